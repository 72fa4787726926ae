"""CPU oracle for the ZeTT embedding-prediction hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``zett_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker / the timed CPU baseline — never
as the thing shipped.

Contents
--------
hypernet_ref.py   numpy fp32 restatement of the hypernetwork forward, written
                  from SURVEY.md Appendix A (reference:
                  hf_hypernet/modeling_hypernet.py:156-267 and the installed
                  transformers RobertaModel math it calls).
retok_ref.c       plain-C restatement of the retokenizer: byte table,
                  BPE merge and Unigram Viterbi (reference call site
                  zett/utils.py:651-689; algorithm = HF ``tokenizers`` 0.22.2
                  ``BPE::tokenize`` / ``Unigram::tokenize``, a third-party Rust
                  dependency that is not under /root/reference).
retok_ref.py      ctypes loader for retok_ref.c plus a pure-Python twin for
                  small cases.

Parity pinning: the reference holds no tests and no golden vectors (SURVEY.md
§4), so the oracle is pinned against outputs of the reference itself run in the
build container: ``tests/golden/make_golden.py`` imports
``/root/reference/hf_hypernet`` and ``zett.utils.get_surface_form_matrix`` and
writes the fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks the oracle against them.
"""
