"""SparseTensor: features F [N, C] + lazily fetched coordinates C [N, D+1] + the key of its
coordinate map + the manager that owns the map (reference:
MinkowskiEngine/MinkowskiSparseTensor.py:48-345 and MinkowskiTensor.py:139-604)."""
import enum
import warnings

import torch

from . import host as _host
from .host import CoordinateMapKey
from .common import convert_to_int_list
from .coordinate_manager import CoordinateManager


class SparseTensorOperationMode(enum.Enum):
    SEPARATE_COORDINATE_MANAGER = 0
    SHARE_COORDINATE_MANAGER = 1


class SparseTensorQuantizationMode(enum.Enum):
    RANDOM_SUBSAMPLE = 0
    UNWEIGHTED_AVERAGE = 1
    UNWEIGHTED_SUM = 2
    NO_QUANTIZATION = 3
    MAX_POOL = 4
    SPLAT_LINEAR_INTERPOLATION = 5


_sparse_tensor_operation_mode = SparseTensorOperationMode.SEPARATE_COORDINATE_MANAGER
_global_coordinate_manager = None


def set_sparse_tensor_operation_mode(operation_mode):
    """MinkowskiTensor.py:71-103"""
    assert isinstance(operation_mode, SparseTensorOperationMode)
    global _sparse_tensor_operation_mode
    _sparse_tensor_operation_mode = operation_mode


def sparse_tensor_operation_mode():
    return _sparse_tensor_operation_mode


def global_coordinate_manager():
    return _global_coordinate_manager


def set_global_coordinate_manager(coordinate_manager):
    global _global_coordinate_manager
    _global_coordinate_manager = coordinate_manager


def clear_global_coordinate_manager():
    set_global_coordinate_manager(None)


class SparseTensor:
    def __init__(self, features, coordinates=None, tensor_stride=1, coordinate_map_key=None,
                 coordinate_manager=None, quantization_mode=SparseTensorQuantizationMode.RANDOM_SUBSAMPLE,
                 allocator_type=None, minkowski_algorithm=None, requires_grad=None, device=None):
        assert isinstance(features, torch.Tensor), "Features must be a torch.Tensor"
        assert features.ndim == 2, f"The feature should be a matrix, The input feature is an order-{features.ndim} tensor."
        assert isinstance(quantization_mode, SparseTensorQuantizationMode)
        self.quantization_mode = quantization_mode
        if coordinates is not None:
            assert isinstance(coordinates, torch.Tensor)
        if coordinate_map_key is not None:
            assert isinstance(coordinate_map_key, CoordinateMapKey)
            assert coordinate_manager is not None, "Must provide coordinate_manager if coordinate_map_key is provided"
            assert coordinates is None, "Must not provide coordinates if coordinate_map_key is provided"
        if coordinate_manager is not None:
            assert isinstance(coordinate_manager, CoordinateManager)
        if coordinates is None and (coordinate_map_key is None or coordinate_manager is None):
            raise ValueError("Either coordinates or (coordinate_map_key, coordinate_manager) pair must be provided.")

        if device is not None:
            features = features.to(device)
            if coordinates is not None:
                coordinates = coordinates.to(device)

        self._D = coordinates.size(1) - 1 if coordinates is not None else coordinate_manager.D
        own_manager = False

        if coordinate_manager is None:
            if _sparse_tensor_operation_mode == SparseTensorOperationMode.SHARE_COORDINATE_MANAGER:
                coordinate_manager = global_coordinate_manager()
                if coordinate_manager is None:
                    coordinate_manager = CoordinateManager(D=self._D, allocator_type=allocator_type,
                                                           minkowski_algorithm=minkowski_algorithm)
                    set_global_coordinate_manager(coordinate_manager)
            else:
                coordinate_manager = CoordinateManager(D=self._D, allocator_type=allocator_type,
                                                       minkowski_algorithm=minkowski_algorithm)
                own_manager = True
        self._manager = coordinate_manager

        if coordinates is not None:
            assert features.shape[0] == coordinates.shape[0], \
                "The number of rows in features and coordinates must match."
            assert features.is_cuda == coordinates.is_cuda, "Features and coordinates must have the same backend."
            coordinate_map_key = _host.key_like(coordinate_manager, convert_to_int_list(tensor_stride, self._D), "")
            coordinates, features, coordinate_map_key = self.initialize_coordinates(
                coordinates, features, coordinate_map_key)
            if own_manager:
                from .coordinate_manager import _prefetch_from_previous
                _prefetch_from_previous(coordinate_manager)
        else:
            assert coordinate_map_key.is_key_set(), "The coordinate key must be valid."

        if requires_grad is not None:
            features.requires_grad_(requires_grad)
        self._F = features
        self._C = coordinates
        self.coordinate_map_key = coordinate_map_key

    # MinkowskiSparseTensor.py:293-345
    def initialize_coordinates(self, coordinates, features, coordinate_map_key):
        if coordinates.dtype != torch.int32:
            warnings.warn("coordinates implicitly converted to torch.IntTensor. To remove this warning, use "
                          "`.int()` to convert the coords into an torch.IntTensor")
            coordinates = torch.floor(coordinates).int()
        coordinates = coordinates.contiguous()
        coordinate_map_key, (unique_index, inverse_mapping) = self._manager.insert_and_map(
            coordinates, *coordinate_map_key.get_key())
        self.unique_index = unique_index.long()
        self.inverse_mapping = inverse_mapping
        n_unique = self.unique_index.numel()
        if n_unique == coordinates.shape[0]:
            # no duplicates: first-occurrence order == input order
            return coordinates, features, coordinate_map_key
        coordinates = coordinates[self.unique_index]
        mode = self.quantization_mode
        if mode in (SparseTensorQuantizationMode.UNWEIGHTED_SUM, SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE):
            # duplicate coordinates: rows of a voxel are summed in input order by one HIP kernel (deterministic;
            # the reference goes through cuSPARSE coo_spmm, MinkowskiSparseTensor.py:317-341)
            from .utils.quantization import segment_reduce
            features = segment_reduce(features, inverse_mapping, n_unique,
                                      average=(mode == SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE))
        elif mode == SparseTensorQuantizationMode.RANDOM_SUBSAMPLE:
            features = features[self.unique_index]
        return coordinates, features, coordinate_map_key

    # ---- accessors (MinkowskiTensor.py:139-330) ---------------------------------------------------
    @property
    def coordinate_key(self):
        return self.coordinate_map_key

    @property
    def coordinate_manager(self):
        return self._manager

    @property
    def tensor_stride(self):
        return self.coordinate_map_key.get_tensor_stride()

    @property
    def C(self):
        if self._C is None:
            self._C = self._manager.get_coordinates(self.coordinate_map_key)
        return self._C

    @property
    def coordinates(self):
        return self.C

    @property
    def F(self):
        return self._F

    @property
    def features(self):
        return self._F

    @property
    def D(self):
        return self._D

    @property
    def dimension(self):
        return self._D

    @property
    def requires_grad(self):
        return self._F.requires_grad

    def requires_grad_(self, requires_grad=True):
        self._F.requires_grad_(requires_grad)
        return self

    @property
    def dtype(self):
        return self._F.dtype

    @property
    def device(self):
        return self._F.device

    @property
    def shape(self):
        return self._F.shape

    def size(self):
        return self._F.size()

    def __len__(self):
        return len(self._F)

    def float(self):
        self._F = self._F.float()
        return self

    def detach(self):
        return SparseTensor(self._F.detach(), coordinate_map_key=self.coordinate_map_key,
                            coordinate_manager=self._manager)

    # ---- arithmetic between tensors that share a coordinate map (MinkowskiTensor.py:390-520) -----
    def _binary(self, other, op):
        if isinstance(other, SparseTensor):
            assert other._manager is self._manager, "coordinate managers must match"
            if self.coordinate_map_key != other.coordinate_map_key:
                raise NotImplementedError("binary operations across different coordinate maps (union maps) are "
                                          "outside the hot path")
            return SparseTensor(op(self._F, other._F), coordinate_map_key=self.coordinate_map_key,
                                coordinate_manager=self._manager)
        return SparseTensor(op(self._F, other), coordinate_map_key=self.coordinate_map_key,
                            coordinate_manager=self._manager)

    def __add__(self, other):
        return self._binary(other, torch.add)

    def __radd__(self, other):
        return self._binary(other, torch.add)

    def __iadd__(self, other):
        return self._binary(other, torch.add)

    def __sub__(self, other):
        return self._binary(other, torch.sub)

    def __mul__(self, other):
        return self._binary(other, torch.mul)

    def __truediv__(self, other):
        return self._binary(other, torch.div)

    def __neg__(self):
        return SparseTensor(-self._F, coordinate_map_key=self.coordinate_map_key, coordinate_manager=self._manager)

    @property
    def _batchwise_row_indices(self):
        batch = self.C[:, 0]
        return [torch.nonzero(batch == b, as_tuple=False).flatten() for b in torch.unique(batch).tolist()]

    @property
    def decomposed_coordinates(self):
        return [self.C[idx, 1:] for idx in self._batchwise_row_indices]

    @property
    def decomposed_features(self):
        return [self._F[idx] for idx in self._batchwise_row_indices]

    def __repr__(self):
        return (f"{self.__class__.__name__}(\n  coordinates={self.C}\n  features={self._F}\n  "
                f"coordinate_map_key={self.coordinate_map_key}\n  coordinate_manager={self._manager}"
                f"  spatial dimension={self._D})")


def _get_coordinate_map_key(input, coordinates=None, tensor_stride=1, expand_coordinates=False):
    """Key of the output map of an operator (MinkowskiSparseTensor.py:754-783)."""
    if coordinates is not None and not expand_coordinates:
        assert isinstance(coordinates, (CoordinateMapKey, torch.Tensor, SparseTensor))
        if isinstance(coordinates, torch.Tensor):
            assert coordinates.ndim == 2
            key = _host.key_like(input._manager, convert_to_int_list(tensor_stride, coordinates.size(1) - 1), "")
            key, _ = input._manager.insert_and_map(coordinates, *key.get_key())
            return key
        if isinstance(coordinates, SparseTensor):
            return coordinates.coordinate_map_key
        return coordinates
    return _host.key_like(input.coordinate_map_key)
