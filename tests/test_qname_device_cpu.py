"""CPU check of the HOST half of the device QNAME path (uq_amd/qname_device.py): the closed form that
replaces the reference's sequential loop (uq.py:394-444) and the checkpointed typing rules (uq.py:586-676),
fed by numpy stand-ins for the kernels (tests/fake_qname_ops.py), against the oracle.  The same cases run
on the real kernels in tests/test_gpu_qname.py."""
import numpy as np
import pytest
import torch

import fake_qname_ops as F
import oracle_c
import test_gpu_qname as T
from uq_amd import qname_device


@pytest.fixture()
def fake(monkeypatch):
    def index(ctx, fq):
        host = np.frombuffer(fq, dtype=np.uint8).copy()
        ls = oracle_c.index_lines(host)
        return torch.from_numpy(host), torch.from_numpy(ls.view(np.int64).copy()), (len(ls) - 1) // 4
    monkeypatch.setattr(T, 'INDEX', index)
    monkeypatch.setattr(T, 'FUSED', False)
    monkeypatch.setattr(qname_device, 'ops', F)
    return F.FakeCtx()


@pytest.mark.parametrize('case', ['test_synthetic_illumina_names', 'test_golden_fastq', 'test_mapping_columns_and_suffix',
                                  'test_long_integer_fields', 'test_demotion_checkpoints', 'test_refusals_and_declines',
                                  'test_random_grammars_differential', 'test_mutated_names_differential'])
def test_host_logic_with_numpy_kernels(fake, case):
    getattr(T, case)(fake)
