function x = vamp(y, A, sigma, L)
% Drop-in for benchmark_algorithms/vamp.m (dense dictionary with at most 128 rows; for the drivers'
% Kronecker dictionary kron((B*B').', A) call the C ABI's jstsp_vamp_kron_c32 with the factors).
  x = jstsp_mex('vamp', y, A, sigma, L);
end
