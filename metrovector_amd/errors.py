"""Error vocabulary of the MVF host side — one exception per `MvfError`
variant of the reference (src/errors.rs:8-40), keyed by the C ABI's status
codes (include/mvf_status.h)."""
from __future__ import annotations


class MvfError(Exception):
    """Base of all MVF errors (reference: enum MvfError, src/errors.rs:8)."""

    status = -1


class IoError(MvfError):  # MvfError::Io
    status = 1


class InvalidFormat(MvfError):  # MvfError::InvalidFormat
    status = 2


class UnsupportedVersion(MvfError):  # MvfError::UnsupportedVersion
    status = 3


class VectorSpaceNotFound(MvfError):  # MvfError::VectorSpaceNotFound
    status = 4


class IndexOutOfBounds(MvfError):  # MvfError::IndexOutOfBounds
    status = 5


class DimensionMismatch(MvfError):  # MvfError::DimensionMismatch
    status = 6


class InvalidVectorType(MvfError):  # MvfError::InvalidVectorType
    status = 7


class CorruptedData(MvfError):  # MvfError::CorruptedData
    status = 8


class ExtensionError(MvfError):  # MvfError::Extension
    status = 9


class BuildError(MvfError):  # MvfError::Build
    status = 10


class DeviceError(MvfError):  # new: HIP / RCCL failure or no GPU
    status = 11


class InvalidArgument(MvfError):  # new
    status = 12


_BY_STATUS = {c.status: c for c in (IoError, InvalidFormat, UnsupportedVersion, VectorSpaceNotFound,
                                    IndexOutOfBounds, DimensionMismatch, InvalidVectorType, CorruptedData,
                                    ExtensionError, BuildError, DeviceError, InvalidArgument)}


def raise_for_status(status: int, detail: str = "") -> None:
    if status == 0:
        return
    cls = _BY_STATUS.get(status, MvfError)
    raise cls(detail or f"status {status}")
