set -o pipefail
mkdir -p gpurun_out/r02r
bash bench/steps.sh gpurun_out/r02r \
 "300 ab python bench/ab.py --spp 32 --rounds 2 new=pathtracer_amd/libhpt.so nt=build/libhpt_nt.so" \
 "300 ab_c4 python bench/ab.py --scene tenm --spp 4 --rounds 2 new=pathtracer_amd/libhpt.so nt=build/libhpt_nt.so" \
 "300 ab_c2 python bench/ab.py --scene bunny --spp 64 --rounds 2 new=pathtracer_amd/libhpt.so nt=build/libhpt_nt.so"
grep -h "^{" gpurun_out/r02r/ab*.log
