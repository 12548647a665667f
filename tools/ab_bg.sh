run() { env "$@" timeout 300 python bench.py --steps 48 --warmup 24 --no-cpu-baseline --no-extras $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(sys.argv[1:], d['value'], {k.split('_')[1]+k.split('_')[2]:(v['frac'],round(v['avg_launch_ms'],2)) for k,v in d['rooflines'].items()}, d['request_latency_ms']['p50'])" "$@"; }
echo "== probe, blit threshold"; for b in 16 0; do echo BLIT=$b; GPU_FORCE_BLIT_COPY_SIZE=$b python tools/single_page_probe.py 12 360; done
echo "== default bench"
run GPU_FORCE_BLIT_COPY_SIZE=16
run GPU_FORCE_BLIT_COPY_SIZE=0
run GPU_FORCE_BLIT_COPY_SIZE=0 OCRS_GX_HEAVY=1
