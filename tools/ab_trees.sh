#!/bin/bash
# same-box A/B of WHOLE TREES (python + library + plan): tools/ab_trees.sh <rounds> <tree under tmp_ab/> -- the tree's own bench.py against the current one,
# interleaved; prints ms_per_step (two lanes) of each run.  The tree is a `git archive` snapshot (tmp_ab/ is git-ignored but travels to the GPU box).
R=$GRAFT_REPO_ROOT
n=$1; t=$2
(cd $R/tmp_ab/$t/scrfd_arcface_facerecognition_amd/csrc && make -j16 > /tmp/build_$t.log 2>&1) || { echo "build $t failed"; tail -3 /tmp/build_$t.log; exit 1; }
one() { (cd $1 && python bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-roofline $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$2', d['ms_per_step'], d.get('ms_per_step_1lane'))"); }
for i in $(seq $n); do one $R/tmp_ab/$t $t; one $R work; done
