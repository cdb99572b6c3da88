"""CPU oracle for the caption-decoder hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  Nothing under ``imagecaption_amd/`` imports it.

Parity status: the reference (/root/reference) ships no tests, golden vectors or checkpoints for
this path, and its own modules cannot be imported here (torchvision / nltk / rouge_score / spacy
are absent; no permission denial was involved -- see SURVEY.md section 8c).  The arithmetic of the
path lives in PyTorch itself, so the oracle is the reference's module arrangement rebuilt from the
same stock ``torch.nn`` classes on CPU (``torch_oracle``), cross-checked by a second restatement
written with plain matmul / softmax / layer-norm (``plain_oracle``).  The only fixture the
reference holds is the vocabulary pickle, which pins the tokenizer restatement
(``oracle/tokenizer.py``), not the decoder: decoder parity is "unpinned by the reference" and
anchored on stock torch.nn, which is what the reference itself calls.
"""
