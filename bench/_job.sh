set -o pipefail
mkdir -p gpurun_out/r02n
bash bench/steps.sh gpurun_out/r02n \
 "300 ab python bench/ab.py --spp 32 --rounds 3 new=pathtracer_amd/libhpt.so r6=build/libhpt_regroup6.so" \
 "300 ab_c4 python bench/ab.py --scene tenm --spp 4 --rounds 2 new=pathtracer_amd/libhpt.so r6=build/libhpt_regroup6.so" \
 "300 ab_c2 python bench/ab.py --scene bunny --spp 64 --rounds 2 new=pathtracer_amd/libhpt.so r6=build/libhpt_regroup6.so" \
 "300 ab_c5 python bench/ab.py --scene glossy --spp 8 --rounds 2 new=pathtracer_amd/libhpt.so r6=build/libhpt_regroup6.so"
grep -h "^{" gpurun_out/r02n/ab*.log
