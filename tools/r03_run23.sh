#!/bin/bash
timeout -k 10 400 python tools/skip_probe.py 2>/dev/null && echo "--- single stream" && O2M_WGRAD_STREAM=0 O2M_GROUP_STREAM=0 O2M_SIDE_STYLE=0 timeout -k 10 400 python tools/skip_probe.py 2>/dev/null
