#!/bin/bash
python tools/host_time.py 2>/dev/null && O2M_WGRAD_STREAM=0 O2M_GROUP_STREAM=0 O2M_SIDE_STYLE=0 python tools/host_time.py 2>/dev/null && SIZE=32 BATCH=2 python tools/host_time.py 2>/dev/null && SIZE=64 BATCH=4 python tools/host_time.py 2>/dev/null
