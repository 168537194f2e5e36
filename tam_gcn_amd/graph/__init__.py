from . import tools
from . import ntu_rgb_d
from . import ucla
from . import synthetic
