# isolated per-class kernel timing of the current build (3 repetitions)
for i in 1 2 3; do
timeout 200 python bench.py --pages 8 --steps 8 --warmup 3 --inflight 1 --no-cpu-baseline --no-extras --profile-hint --no-pipeline 2>&1 >/dev/null | grep -E "gemm_conv3x3|gemm_gru_hidden|gemm_gru_input"
done
