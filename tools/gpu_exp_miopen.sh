#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python tools/exp_extract2.py 2>&1 | grep MIOPEN
MIOPEN_FIND_MODE=1 timeout 600 python tools/exp_extract2.py 2>&1 | grep MIOPEN
MIOPEN_FIND_MODE=3 timeout 600 python tools/exp_extract2.py 2>&1 | grep MIOPEN
MIOPEN_FIND_MODE=1 MIOPEN_FIND_ENFORCE=3 timeout 1500 python tools/exp_extract2.py 2>&1 | grep MIOPEN
ls ~/.config/miopen 2>/dev/null | head; du -sh ~/.config/miopen 2>/dev/null
