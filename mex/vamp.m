function x = vamp(y, A, sigma, L)
% Drop-in for benchmark_algorithms/vamp.m (dense dictionary, min(size(A)) <= 2048: the drivers' call with
% Phi = kron((B*B').', A), 512 x 512, goes through as it is; jstsp_mex('vamp_kron', Y*B', A, B*B', sigma, L) is the
% same estimate from the factors, without forming Phi).
  x = jstsp_mex('vamp', y, A, sigma, L);
end
