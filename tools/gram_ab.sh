#!/bin/bash
# A/B of gram kernel variants with the order alternated (the GPU needs ~20 ms of work to reach its steady clock, so a fixed order
# confounds the comparison): tools/gram_ab.sh "PMT_GRAM_SK_ABLATE=0" "PMT_GRAM_SK_ABLATE=4" ...
for rep in 1 2 3; do
  for v in "$@"; do
    echo "[$rep] $v: $(env $v python tools/gram_probe.py 4096x4096 16384x4096 2>&1 | grep rows | tr '\n' ' ')"
  done
done
