"""A/B inside one process launch sequence: bench.py with the K7 -> K8 scratch kept clean between calls (the product path,
GsrGrads.scratch_clean = 1) against the protocol without the contract (the library clears the sums before K7; K8 finds
the reached Gaussians by reading them). usage: python tools/ab_scratch.py legacy|clean [bench.py arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv.pop(1)
from dreamscene_amd import rasterizer as R
if mode == "legacy":
    def _legacy(gr, sc, k=0):
        gr.partials = sc.partials[k].data_ptr()
        gr.reach = None
        gr.scratch_clean = 0
    R._bind_scratch = _legacy
import bench
bench.main()
