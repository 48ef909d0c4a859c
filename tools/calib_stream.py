"""Known byte count for the FETCH_SIZE / WRITE_SIZE calibration: 4 x (1 GiB read + 1 GiB write), 8 B/lane coalesced."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teb_local_planner_amd import planner, scenes
s = planner.make_solver(*scenes.scene_c1())
s.debug_stream((1 << 30) // 8, 4)
s.close()
