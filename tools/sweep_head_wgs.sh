for v in 0 2 3 4 6 8; do
  echo "HEAD_WGS_PER_CU=$v  C5:"; UNETPP_HEAD_WGS_PER_CU=$v DTYPE=bf16 B=4 SIZE=384 CH=64 NCLS=5 REPS=20 python tools/bench_heads.py
  echo "HEAD_WGS_PER_CU=$v  C3:"; UNETPP_HEAD_WGS_PER_CU=$v DTYPE=bf16 B=8 SIZE=512 CH=32 NCLS=4 REPS=20 python tools/bench_heads.py
done
