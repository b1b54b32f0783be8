#!/bin/sh
# Regenerates tests/golden/rng_goldens.json from the REFERENCE's own
# include/caffe/data_generation/SimpleRandom.h (compiled where it lies under
# /root/reference; only possible in the build container).
set -e
cd "$(dirname "$0")/../.."
make -C oracle ref
./oracle/_ref/ref_rng > tests/golden/rng_goldens.json
