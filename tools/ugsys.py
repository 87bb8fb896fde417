#!/usr/bin/env python3
"""`ugsys.py`, the name the reference's README uses for its driver (README.md:39);
the file itself is `hgsys.py`.  Both spellings run the same driver here."""
import os
import runpy

runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hgsys.py"), run_name="__main__")
