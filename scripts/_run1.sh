cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "large_graph_training or big" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "thin_output_head" 2>&1 | tail -2
python scripts/_dbg_b6.py 2>&1 | grep -v amdgpu.ids
echo "--- bf16x6"; python scripts/train_big.py 1e6 1e7 64 10 2>&1 | tail -4
echo "--- f32"; GNN_TRAIN_BF16X6=0 python scripts/train_big.py 1e6 1e7 64 10 2>&1 | tail -3
