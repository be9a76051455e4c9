"""bird-sound-event-detecion_amd: MI355X-native mel + CRNN train-step hot path.

Host-side mirror of the reference's Python API (CRNN / Predictor / preprocess / train step /
update_ema_variables / get_predictions) over the C ABI of ``libbsed.so`` (include/bsed.h).
Import it as ``bsed_amd``.  There is NO CPU fallback: every op raises if the HIP library or a GPU
is missing.
"""
__version__ = "0.1.0"
