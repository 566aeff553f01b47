#!/usr/bin/env python3
"""Drop-in for the reference's `python main_3dhp.py ... --evaluate FILE -num_proposals H -sampling_timesteps K`
(README.md, MPI-INF-3DHP section) on MI355X.  See d3dp_amd/cli.py::main_3dhp."""
import sys

from d3dp_amd.cli import main_3dhp

if __name__ == "__main__":
    sys.exit(main_3dhp())
