import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import bench_neumf as b
b.run(int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 10, bf16=len(sys.argv) > 2)
