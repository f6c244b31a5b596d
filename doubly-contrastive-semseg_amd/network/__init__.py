"""Drop-in for the reference's ``network`` package on the train-step path
(network/__init__.py:1-5): put this directory in front of the reference root on
PYTHONPATH and ``main.py`` picks up the MI355X implementation unchanged."""
from dcs_amd.model import WeatherNet, WeatherClassifier     # noqa: F401
from . import modeling                                      # noqa: F401,E402
