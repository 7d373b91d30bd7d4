"""Drop-in `InkLayer` package for the detector -> segmentor hot path, backed by the MI355X HIP
engines in `inklayer_amd`.  Same module and function names as the reference package
(ooowedyn/InkLayer: InkLayer/runner.py, detector/gdino.py, segmentor/sam.py, utils/processing.py),
so the reference's `main.py` and `custom_interface/app.py` import it unchanged.

Out of scope here: refinement, inpainting, visualisation polish, the Flask UI."""
