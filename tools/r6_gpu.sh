#!/bin/bash
cd /root/repo
timeout -k 5 300 python tools/exp/life_host_prof.py 2>&1 | tail -40
