#!/bin/bash
# Twelve runs of tests/_dp_capture_probe.py (the captured data-parallel step on a one-rank RCCL group) in a row: the capture /
# watchdog race of DESIGN 11.5 aborted about one run in four before TrainStep.capture waited for the watchdog.
cd $GRAFT_REPO_ROOT
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  python tests/_dp_capture_probe.py $((29500+i)) > /tmp/p$i.out 2> /tmp/p$i.err; rc=$?; echo "run $i rc=$rc json=$(grep -c '^{' /tmp/p$i.out)"
  if [ $rc -ne 0 ]; then grep -v Warning /tmp/p$i.err | head -c 2500; echo; echo ---- ; tail -5 /tmp/p$i.out | cut -c1-300; fi
done
