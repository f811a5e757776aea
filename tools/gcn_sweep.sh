for r in 6144 12288 24576 49152 98304 196608 393216; do PROBE_ROWS=$r python tools/gcn_layer_probe.py 2>&1 | grep "\"nodes\": 32, \"graphs\": [0-9]*, \"K\": 200, \"C\": 200" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); rows=d['nodes']*d['graphs']; print(rows, d['fused_mfma_us'], round(d['fused_mfma_us']/ (rows/128) * 256,2), 'us per tile-slot', d['fused_TFLOPs'])"; done
