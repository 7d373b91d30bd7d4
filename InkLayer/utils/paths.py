"""InkLayer/utils/paths.py:4-6 of the reference: checkpoints live in <repo>/models/."""
import os

import InkLayer


def get_model_path(filename):
    return os.path.join(os.path.dirname(InkLayer.__file__), "..", "models", filename)
