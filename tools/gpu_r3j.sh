#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/prof_stream.py 2>/dev/null | grep "rb_\|sum"
