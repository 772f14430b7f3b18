import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_agent_gpu as t
bad = 0
for i in range(12):
    try:
        t.test_graph_replayed_update_equals_eager_update()
    except AssertionError as e:
        bad += 1
        print(i, "FAIL", str(e)[:200], flush=True)
print("bad", bad)
