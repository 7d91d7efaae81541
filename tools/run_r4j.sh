#!/bin/bash
for m in none empty1 empty4k stream8 stream64; do timeout 300 python tools/exp_interference.py $m 2>&1 | tail -1; done
