for tail in 8 16 24 32; do TAIL=$tail python profiles/exp_launch_size.py 2>&1 | grep -v amdgpu | grep -v "rows from"; done
