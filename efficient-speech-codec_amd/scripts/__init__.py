"""CLI counterparts of the reference's scripts/ (compress, test, metrics) that run without torchaudio / pesq."""
