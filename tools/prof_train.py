import os, sys, time, cProfile, pstats
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import deepstructuredmixtures_amd as dsm
model, X, y, Xt, ptr, idx = bench.build_model("dsmgp_n100k_d8", 0, 1, 0)
dsm.train(model, dsm.ADAM(), iterations=1)
pr = cProfile.Profile(); pr.enable()
dsm.train(model, dsm.ADAM(), iterations=3)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
