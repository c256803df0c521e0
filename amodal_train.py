#!/usr/bin/env python
"""Reference entry point name kept at the repository root (amodal_train.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sln_amodal_amd.amodal_train import main  # noqa: E402

if __name__ == "__main__":
    main()
