set -o pipefail
mkdir -p gpurun_out/r02z
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r02z/gpu_tests.log 2>&1
rc=$?; tail -n 3 gpurun_out/r02z/gpu_tests.log
case $rc in 0) ;; *) echo "tests rc=$rc"; exit $rc;; esac
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash bench/profile_all.sh r02z
