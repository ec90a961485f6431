"""Driver for profiling the create_proof replay from any working directory: tools/replay_probe.py [word_bits] [random|witness]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_ram_halo2_amd import replay
replay.run(int(sys.argv[1]) if len(sys.argv) > 1 else 32, columns=sys.argv[2] if len(sys.argv) > 2 else "witness", keygen=False, gates_dir=os.path.join(ROOT, "tests", "golden"))
