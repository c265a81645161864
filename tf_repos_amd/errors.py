"""tf.errors-compatible exception classes raised by the engine (SURVEY 8b 'errors').

TF raises InvalidArgumentError for an out-of-range gather id or a malformed number in
string_to_number, OutOfRangeError at end of input (DeepFM.py:84-96 via tf.data) [TF-1.4].
"""


class OpError(Exception):
    def __init__(self, message="", node_def=None, op=None):
        super().__init__(message)
        self.message = message


class InvalidArgumentError(OpError):
    pass


class OutOfRangeError(OpError):
    pass


class NotFoundError(OpError):
    pass


class InternalError(OpError):
    pass


class UnimplementedError(OpError):
    pass


class DataLossError(OpError):
    """Unrecoverable data loss or corruption (tf.errors.DataLossError): a TFRecord or checkpoint checksum that does not match,
    a truncated record."""
