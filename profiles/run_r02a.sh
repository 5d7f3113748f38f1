cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02a_pytest.txt
bash profiles/collect.sh r02a_triplane_R1 triplane_R1 "ngf::render_kernel" > /dev/null 2>&1
bash profiles/collect.sh r02a_infoinv_R1 infoinv_R1 "ngf::render_kernel" > /dev/null 2>&1
bash profiles/collect.sh r02a_uv_sphere uv_sphere "uv_render_kernel" > /dev/null 2>&1
cat gpurun_out/r02a_pytest.txt
tail -12 gpurun_out/r02a_uv_sphere_pmc.txt
