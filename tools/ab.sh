#!/bin/bash
# usage: tools/ab.sh <command...>: run the command with the current library, then with tools/ab_old/libbnn_hip.so in its place
cd $GRAFT_REPO_ROOT
echo "== new"; "$@"
cp bayesianneuralnetworks_amd/libbnn_hip.so /tmp/new.so
cp tools/ab_old/libbnn_hip.so bayesianneuralnetworks_amd/libbnn_hip.so
echo "== old"; "$@"
cp /tmp/new.so bayesianneuralnetworks_amd/libbnn_hip.so
echo "== new again"; "$@"
