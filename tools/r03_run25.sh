#!/bin/bash
timeout -k 10 300 python tools/pointwise_bw.py 2>/dev/null
