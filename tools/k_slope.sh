set -e
for K in 64 256 512 1024 2048 4096; do
  for tile in "128,128,1,0" "128,128,1,32"; do
    GEMM_DTYPE=bf16s GEMM_TILE=$tile timeout -k 10 60 python tools/gemm_one.py 0 1 5120 1024 $K 30
  done
done
for K in 64 1024 4096; do
    GEMM_DTYPE=bf16s GEMM_TILE=128,128,1,0 timeout -k 10 60 python tools/gemm_one.py 0 1 4096 1024 $K 30
    GEMM_DTYPE=bf16s GEMM_TILE=128,128,1,0 timeout -k 10 60 python tools/gemm_one.py 0 1 8192 1024 $K 30
    GEMM_DTYPE=bf16s GEMM_TILE=128,128,1,0 timeout -k 10 60 python tools/gemm_one.py 0 1 16384 1024 $K 30
done
