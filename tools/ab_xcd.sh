cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
for sh in "sift 8192 8192 50" "surf 8192 8192 20" "orb 32768 32768 10" "sift 32768 32768 10" "sift 4096 4096 50" "orb 8192 8192 30"; do
  timeout -k 10 300 python tools/ab_options.py 8 1,2 $sh 2>&1 | grep -v amdgpu.ids || exit 1
done
