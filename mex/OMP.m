function [x_hat, indexSet, v, targetMatrix] = OMP(A, v, m, snr)
% Drop-in for benchmark_algorithms/OMP.m (snr is unused there too).
  [x_hat, indexSet, v, targetMatrix] = jstsp_mex('OMP', A, v, m, snr);
end
