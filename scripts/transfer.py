#!/usr/bin/env python
"""Drop-in for the reference's scripts/transfer.py (same flags); see zett_amd/transfer.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from zett_amd.transfer import main  # noqa: E402

if __name__ == "__main__":
    main()
