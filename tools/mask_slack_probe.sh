#!/bin/bash
# one process per setting: CTK_MASK_SLACK_MB is read once.  Output: gpurun_out/slack/probe.txt
mkdir -p gpurun_out/slack
export CTK_MASK_TUNE=0
for rep in 1 2; do
for mb in 0 64 512 2048 8192; do
  if [ $mb = 0 ]; then unset CTK_MASK_SLACK_MB; else export CTK_MASK_SLACK_MB=$mb; fi
  python tools/mask_slack_probe.py
done
done > gpurun_out/slack/probe.txt 2>&1
unset CTK_MASK_SLACK_MB
SHAPE=480,721,1440 python tools/mask_slack_probe.py >> gpurun_out/slack/probe.txt 2>&1
CTK_MASK_SLACK_MB=2048 SHAPE=480,721,1440 python tools/mask_slack_probe.py >> gpurun_out/slack/probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/slack/avail.txt 2>&1
