#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/r06/phase_small.py 2>&1 | grep -v amdgpu | tail -40
