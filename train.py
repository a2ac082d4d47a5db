#!/usr/bin/env python3
"""`python train.py ...` -- same entry point and flags as the reference's train.py; see iswm_amd/train.py."""
from iswm_amd.train import main

if __name__ == "__main__":
    main()
