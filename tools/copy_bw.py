import sys; sys.path.insert(0,'.')
from tf_repos_amd.engine import Engine
import torch
torch.cuda.set_device(0)
for nb in (1<<28, 1<<30, 1<<31):
    print(nb>>20, "MiB", [round(Engine.measure_copy_bandwidth(nb, 20),1) for _ in range(3)])
