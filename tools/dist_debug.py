import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch.multiprocessing as mp
from dist_workers import engine_sharded_solve
if __name__ == "__main__":
    d = tempfile.mkdtemp()
    mp.spawn(engine_sharded_solve, args=(2, 29641, d), nprocs=2, join=True)
    print("done", os.listdir(d))
