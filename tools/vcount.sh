#!/bin/bash
# Static count of the vector instructions of k_render<0> in a compiled assembly file (tools/build/pt_hip.s by default):
# total, register copies, and by class. A quick metric for source-level experiments (time = vector instructions issued).
f=${1:-tools/build/pt_hip.s}
awk '/^_Z8k_renderILi0EEv12RenderParams:/,/s_endpgm/' $f > /tmp/kr0.s
echo "vector instructions: $(grep -c '^\s*v_' /tmp/kr0.s)  v_mov: $(grep -c '^\s*v_mov' /tmp/kr0.s)  cndmask: $(grep -c '^\s*v_cndmask' /tmp/kr0.s)  readlane/writelane: $(grep -c 'v_readlane\|v_writelane' /tmp/kr0.s)  s_nop: $(grep -c 's_nop' /tmp/kr0.s)"
