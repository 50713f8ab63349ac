#!/usr/bin/env python3
"""Run the 8(f) timing tools in ONE process so a single rocprofv3 trace covers
all of their kernels (tools/collect_profiles.sh uses this)."""
import os
import runpy
import sys

here = os.path.dirname(os.path.abspath(__file__))
for tool, argv in (("tf_time.py", []), ("tf_time.py", ["--bd", "10"]),
                   ("compound_time.py", []), ("wiener_time.py", []), ("hbd_time.py", ["--bd", "8"]), ("hbd_time.py", ["--bd", "10"])):
    sys.argv = [tool] + argv
    runpy.run_path(os.path.join(here, tool), run_name="__main__")
