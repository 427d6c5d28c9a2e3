#!/bin/bash
# Root of the input data set, laid out like the reference's genarch-inputs
# (<bench>/small, <bench>/large; /root/reference/README.md:5-44).  tests/make_inputs.py builds such a tree
# from the seeded generators when the original data set is not available.
export GENARCH_BENCH_INPUTS_ROOT="PATH_TO/genarch-inputs"
