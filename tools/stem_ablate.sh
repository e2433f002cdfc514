#!/bin/bash
# phase ablations of scrfd_stem_rows (FID_STEM_ABLATE: 1 conv0, 2 conv1, 4 conv2 + pool, 8 stores, 16 input; wrong results), us per 64 frames
for e in 0 1 2 4 8 16 3 7 15 23 31 6 0; do
  echo "[$e] $(FID_STEM_ABLATE=$e python3 tools/run_stem.py 64 40 2>/dev/null | tail -1)"
done
