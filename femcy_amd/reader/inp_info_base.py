"""abstract reader interface (same method set as /root/reference/reader/inp_info_base.py:12-40)."""
import abc


class InpInfoBase(abc.ABC):
    """what a mesh/BC/material reader must provide to System_of_equations.solve()."""

    @abc.abstractmethod
    def __init__(self, file_name: str): ...

    @abc.abstractmethod
    def read_node_element(self, file_name: str): ...

    @abc.abstractmethod
    def read_set(self, file_name: str): ...

    @abc.abstractmethod
    def read_face_set(self, file_name: str): ...

    @abc.abstractmethod
    def get_boundary_condition(self, file_name: str): ...

    @abc.abstractmethod
    def read_material(self, file_name: str): ...

    @abc.abstractmethod
    def read_geometric_nonlinear(self, file_name: str): ...

    @abc.abstractmethod
    def read_time_inc(self, file_name: str): ...
