#!/usr/bin/env python3
"""Drop-in for the reference's `python main.py ... --evaluate FILE -num_proposals H -sampling_timesteps K -b B`
(README.md:39) on MI355X.  See d3dp_amd/cli.py."""
import sys

from d3dp_amd.cli import main

if __name__ == "__main__":
    sys.exit(main())
