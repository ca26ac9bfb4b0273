"""Repo-root shim so that ``python -m cfl.bin.<x>`` (the command lines of the
reference's experiments/*/run.sh and eval.sh) works from the repository root: the
real package lives in compatibility-family-learning_amd/cfl/."""
import os

__path__ = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                         'compatibility-family-learning_amd', 'cfl')]
