cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_registration.py tests/test_gpu_golden.py "tests/test_gpu_properties.py::test_a6_closed_form_sensitivity_at_full_size" tests/test_oracle_golden.py tests/test_oracle_vs_ref.py -q -x 2>&1 | tail -15
