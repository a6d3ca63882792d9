cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --batch 4 --steps 80 --warmup 15 --no-cpu-baseline --no-strict --profile-steps 0 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench dist', d['ms_per_step'], 'host loop', d['config']['host_loop_ms_per_step'])"
python bench.py --batch 4 --steps 80 --warmup 15 --no-cpu-baseline --no-strict --profile-steps 0 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench plain', d['ms_per_step'], 'host loop', d['config']['host_loop_ms_per_step'])"
MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python - <<'PY'
import os, sys, time, cProfile, pstats
sys.argv = ["bench.py", "--batch", "4", "--steps", "200", "--warmup", "15", "--no-cpu-baseline", "--no-strict", "--profile-steps", "0"]
import bench
pr = cProfile.Profile(); pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime")
rows = sorted(((tt, nc, ct, f) for f, (cc, nc, tt, ct, callers) in st.stats.items()), reverse=True)[:25]
for tt, nc, ct, f in rows:
    print(f"{nc:8d} calls own {tt*1e3:9.1f} ms cum {ct*1e3:9.1f} ms {os.path.basename(f[0])}:{f[1]} {f[2]}")
PY
