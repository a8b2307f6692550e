"""ipc_amd -- MI355X-native consistency-matrix / consensus-maximisation engine for IPC."""
