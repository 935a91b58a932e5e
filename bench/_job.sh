set -o pipefail
mkdir -p gpurun_out/r02t
bash bench/steps.sh gpurun_out/r02t \
 "400 ab python bench/ab.py --spp 32 --rounds 3 base=pathtracer_amd/libhpt.so camlast=build/libhpt_camlast.so gray=build/libhpt_gray.so"
grep -h "^{" gpurun_out/r02t/ab*.log
