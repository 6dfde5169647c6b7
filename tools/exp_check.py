#!/usr/bin/env python3
"""Run pytest against an experiment build of the library (tools/build_variant.sh NAME ...): tools/exp_check.py NAME [pytest args]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import models  # noqa
from scalable_e3_gnn_amd import _lib as _L
_L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), "exp", f"libe3gnn_{sys.argv[1]}.so")
import pytest
sys.exit(pytest.main(sys.argv[2:]))
