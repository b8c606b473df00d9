#!/bin/bash
cd tools && timeout -k 10 300 python bench_p8_fit.py 2>/dev/null
