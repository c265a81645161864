import sys, json
sys.path.insert(0, '.')
import bench
r = bench.end_to_end(dict(bench.CONFIGS['c2']))
print(json.dumps({k: r[k] for k in ('examples_per_sec', 'ms_per_step', 'steps', 'wall_s')}))
