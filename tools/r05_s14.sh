#!/bin/bash
# round 5, GPU session 14: one launch vs two concurrent half-batch launches
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python tools/ab_split.py 2>&1 | grep -v "^/opt"
