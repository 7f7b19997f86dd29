# developer A/B of the crowded-block hand-over (FLIMO_HEAVY=<n>, default off): bench value, interleaved repeats
for rep in 1 2 3; do for h in 0 128 192; do FLIMO_HEAVY=$h python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('heavy $h bench', round(d['value']), round(d['ms_per_step']*1e3,1), round(d['roofline']['mean_launch_us'],2))"; done; done
