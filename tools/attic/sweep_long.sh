# needs a tuning build of the library: make -C gaussiananything_amd/csrc clean all EXTRA=-DGA_TUNING
for l in 9 10 11 12 30; do echo "LONG_LOG2=$l"; GA_LONG_LOG2=$l python bench.py --no-cpu-baseline --no-dit | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'])"; done
GA_LONG_LOG2=10 python bench.py --no-cpu-baseline --no-dit --scene stress | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stress', d['ms_per_step'], d['stage_ms'])"
