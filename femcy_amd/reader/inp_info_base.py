"""reader interface: what `System_of_equations.solve` and `main.py` expect from any mesh / boundary-condition /
material reader (the role of /root/reference/reader/inp_info_base.py).  A concrete reader must provide every method
named in `REQUIRED`; the check happens when the subclass is defined, not when it is instantiated."""

REQUIRED = ("read_node_element",        # -> nodes f64[nn, dm], {abaqus_type: connectivity}
            "read_set",                 # -> node sets, element sets (0-based index arrays)
            "read_face_set",            # -> {surface name: set of sorted global-node tuples}
            "get_boundary_condition",   # -> Dirichlet list, Neumann list
            "read_material",            # -> {keyword: material_zoo object}
            "read_geometric_nonlinear", # -> bool (nlgeom)
            "read_time_inc")            # -> {"ini_inc", "max_time", "min_inc", "max_inc"}


class InpInfoBase:
    #: attributes a reader instance exposes after construction
    ATTRIBUTES = ("nodes", "eSets", "ELE", "node_sets", "ele_sets", "face_sets", "dirichlet_bc_info",
                  "neumann_bc_info", "materials", "geometric_nonlinear", "time_incs")

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        missing = [m for m in REQUIRED if not callable(getattr(cls, m, None))]
        if missing:
            raise TypeError(f"{cls.__name__} does not implement the reader interface: missing {missing}")

    def __init__(self, file_name: str):
        raise NotImplementedError("InpInfoBase is an interface; use reader.InpInfo")
