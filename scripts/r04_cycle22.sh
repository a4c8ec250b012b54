#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 700 python -m pytest tests -m gpu -q -x 2>&1 | tail -4; done
