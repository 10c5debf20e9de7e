import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from ppbo_hartmann6 import run
run(queries=3)          # warm
pr = cProfile.Profile(); pr.enable()
run(queries=20)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(22)
