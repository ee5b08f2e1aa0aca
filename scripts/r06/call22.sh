#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | cut -c1-300
