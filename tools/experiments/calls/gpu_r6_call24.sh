python tools/cull_crossover.py outliers
python tools/cull_crossover.py quick
