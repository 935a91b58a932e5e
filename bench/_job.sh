set -o pipefail
mkdir -p gpurun_out/r02s
bash bench/steps.sh gpurun_out/r02s \
 "200 parity HPT_LEAF_ALIGN=1 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 --timeout-method thread -k 'bit_exact'" \
 "300 ab python bench/ab.py --spp 32 --rounds 2 base=pathtracer_amd/libhpt.so align=pathtracer_amd/libhpt.so,HPT_LEAF_ALIGN=1" \
 "300 ab_c2 python bench/ab.py --scene bunny --spp 64 --rounds 2 base=pathtracer_amd/libhpt.so align=pathtracer_amd/libhpt.so,HPT_LEAF_ALIGN=1" \
 "300 ab_c4 python bench/ab.py --scene tenm --spp 4 --rounds 2 base=pathtracer_amd/libhpt.so align=pathtracer_amd/libhpt.so,HPT_LEAF_ALIGN=1"
grep -h "^{" gpurun_out/r02s/ab*.log; tail -2 gpurun_out/r02s/parity.log
