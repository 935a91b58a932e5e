set -o pipefail
mkdir -p gpurun_out/r02v
bash bench/steps.sh gpurun_out/r02v \
 "300 parity python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 --timeout-method thread -k 'iter or bounce or live or golden'" \
 "120 iter_old HPT_LIB=build/libhpt_regroup6.so python bench/iter_rate.py sponza 128" \
 "120 iter_new python bench/iter_rate.py sponza 128" \
 "120 iter_old2 HPT_LIB=build/libhpt_regroup6.so python bench/iter_rate.py sponza 128" \
 "120 iter_new2 python bench/iter_rate.py sponza 128" \
 "120 iter_old_c4 HPT_LIB=build/libhpt_regroup6.so python bench/iter_rate.py tenm 32" \
 "120 iter_new_c4 python bench/iter_rate.py tenm 32"
