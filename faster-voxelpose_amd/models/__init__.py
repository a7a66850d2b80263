"""Mirror of the reference's ``lib/models`` package namespace (lib/models/__init__.py)."""
from . import faster_voxelpose  # noqa: F401
from . import human_detection_net  # noqa: F401
from . import joint_localization_net  # noqa: F401
from . import project_whole  # noqa: F401
from . import project_individual  # noqa: F401
from . import cnns_2d  # noqa: F401
from . import cnns_1d  # noqa: F401
from . import weight_net  # noqa: F401
