set -o pipefail
mkdir -p gpurun_out/r02u
bash bench/steps.sh gpurun_out/r02u \
 "200 parity HPT_LIB=build/libhpt_morton.so python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 --timeout-method thread -k 'bit_exact and not shard'" \
 "400 ab python bench/ab.py --spp 32 --rounds 3 base=pathtracer_amd/libhpt.so morton=build/libhpt_morton.so" \
 "300 ab_c2 python bench/ab.py --scene bunny --spp 64 --rounds 2 base=pathtracer_amd/libhpt.so morton=build/libhpt_morton.so" \
 "300 ab_c1 python bench/ab.py --scene cornell --spp 64 --rounds 2 base=pathtracer_amd/libhpt.so morton=build/libhpt_morton.so"
grep -h "^{" gpurun_out/r02u/ab*.log; tail -2 gpurun_out/r02u/parity.log
