set -e
mkdir -p gpurun_out/r05q
for w in 2048 1792 1536 1280 1024 4096; do
  echo "## GPC_HIP_FUSE_WGS=$w" >> gpurun_out/r05q/sweep.txt
  GPC_HIP_FUSE_WGS=$w python tools/batch_sweep.py 32 64 | cut -c1-230 >> gpurun_out/r05q/sweep.txt
done
echo "## GPC_HIP_NO_FUSE=1" >> gpurun_out/r05q/sweep.txt
GPC_HIP_NO_FUSE=1 python tools/batch_sweep.py 32 64 | cut -c1-260 >> gpurun_out/r05q/sweep.txt
