for a in "--pairs 256 --steps 20 --warmup 5" "--pairs 128 --steps 40 --warmup 10" "--pairs 256 --steps 100" "--pairs 128 --steps 200" "--pairs 64 --steps 400" "--pairs 256 --steps 20 --warmup 5" "--pairs 128 --steps 40 --warmup 10" "--pairs 64 --steps 80 --warmup 20"; do
  RVO_BENCH_NOPROF=1 timeout -k 10 150 python bench.py --no-cpu --no-legs $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', round(d['value']), round(d['ms_per_step'],4))" || exit 1
done
