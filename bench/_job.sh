set -o pipefail
mkdir -p gpurun_out/r02q
bash bench/steps.sh gpurun_out/r02q \
 "200 prof HPT_LIB=build/libhpt_prof.so python bench/step_profile.py sponza 2" \
 "200 prof_c4 HPT_LIB=build/libhpt_prof.so python bench/step_profile.py tenm 1"
