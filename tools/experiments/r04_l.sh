bash tools/experiments/r04_j.sh > /dev/null 2>&1
bash tools/experiments/r04_k.sh > /dev/null 2>&1
tail -12 gpurun_out/r04/j_patch_ab.log; grep -c passed gpurun_out/r04/j_patch_test.log; tail -2 gpurun_out/r04/j_patch_test.log; grep -A3 "PATCH mode" gpurun_out/r04/k_conv_stamps.log
