"""CPU oracle for the ppvector hot path — TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (NumPy + PyTorch-CPU fp32), the arithmetic of the
reference's hot path (yeyupiaoling/VoiceprintRecognition-PaddlePaddle, ``ppvector`` 1.1.1):

    Kaldi Fbank + CMN  ->  ECAPA-TDNN / TDNN forward  ->  cosine head  ->  AAMLoss,
    plus EER / minDCF scoring.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker.  The product package
(``voiceprintrecognition-paddlepaddle_amd/ppvector``) never imports it and has no CPU
fallback: it raises when the HIP library is missing.

Pinning status
--------------
* The reference publishes no golden vectors or tests, and its arithmetic
  lives in three un-vendored third-party packages (PaddlePaddle, paddleaudio, yeaudio) that
  cannot be installed here.  Against the *PaddlePaddle binary* this oracle is therefore
  **parity unpinned**.
* What IS pinned (tests/test_oracle_*.py, tests/golden/):
  - the model graphs (EcapaTdnn, TDNN, SpeakerIdentification, AAMLoss, pooling) are checked
    against the reference's OWN model source files executed from /root/reference through
    ``oracle/paddle_shim`` (a minimal ``paddle`` API restated on torch-CPU; the shim carries the
    third-party op semantics, the reference files carry the graph).  The outputs are frozen
    under tests/golden/ by ``oracle/gen_golden.py``;
  - parameter inventory against the ``paddle.summary`` printout in the reference README
    (README.md:303-351);
  - Fbank against an independent float64 derivation and against the Kaldi-mel helpers in
    ``transformers.audio_utils``.
"""
