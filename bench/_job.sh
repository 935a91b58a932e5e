set -o pipefail
mkdir -p gpurun_out/r02p
bash bench/steps.sh gpurun_out/r02p \
 "300 ab python bench/ab.py --spp 32 --rounds 3 new=pathtracer_amd/libhpt.so narrow=build/libhpt_narrow.so" \
 "300 ab_c2 python bench/ab.py --scene bunny --spp 64 --rounds 2 new=pathtracer_amd/libhpt.so narrow=build/libhpt_narrow.so"
grep -h "^{" gpurun_out/r02p/ab*.log
