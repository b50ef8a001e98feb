#!/usr/bin/env python3
"""developer helper: run another tools/ script against an alternative build of the library.  usage: with_lib.py <lib.so> <script.py> [args]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgpi_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
