"""histogram of definition sizes (flat-AST nodes per definition) by the size classes of the node-parallel lowering"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import opus_amd
from opus_amd import ir_view as V
import oracle_lib as O

nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 64_000_000
src = O.generate(3, nbytes, flags=1)
ctx = opus_amd.Context(0)
u = ctx.get_definitions(src)
defs = u.array(V.A_DEFINITIONS).astype(np.int64)
n_nodes = u.counts()["nodes"]
sizes = np.diff(np.append(defs, n_nodes))
nodes = u.array(V.A_NODES)
kinds = nodes["kind"][defs]
print("definitions", len(defs), "nodes", n_nodes)
edges = [0, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 1 << 30]
for lo, hi in zip(edges, edges[1:]):
    m = (sizes >= lo) & (sizes < hi)
    print(f"[{lo:5d},{hi:10d})  defs {m.sum():8d}  nodes {sizes[m].sum():10d}  share {sizes[m].sum() / n_nodes:.3f}")
