"""Root-level pytest configuration: `pytest`, `pytest -m gpu` and `pytest -m "not gpu"` from the repository root collect tests/ only
(tools/ holds measurement scripts and shelved experiments, gpurun_out/ the GPU box's scratch output)."""

collect_ignore_glob = ['tools/*', 'gpurun_out/*', 'profiles/*', 'examples/*', 'oracle/*', 'er3t_amd/*', 'bench.py', '__graft_entry__.py']


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'real_clock: mcarats_ng seeds its jobs from the wall clock, as in production (tests/conftest.py freezes it otherwise)')
