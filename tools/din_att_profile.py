"""DIN with attention pooling at the Ali-CCP shape under rocprofv3 (kernel stats): python tools/din_att_profile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "12", "4096"]
os.environ["DCTR_MH_ONLY"] = "din_att:touched_rows"
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "multihot_bench.py")).read())
