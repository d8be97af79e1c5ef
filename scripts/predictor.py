"""Counterpart of the reference's src/predictor.py (same hard-coded surface; see 4dflownet_amd/predictor.py:main)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    importlib.import_module("4dflownet_amd.predictor").main()
