function [S, Y, convergence_error] = proposed_algorithm_angles(subY, Omega, indx_S, A, B, Imax, tau_Y, tau_S, rho, type, greedy_nnz)
% Drop-in for basic_system_functions/proposed_algorithm_angles.m (greedy_nnz is unused there too).
  if nargout >= 3
    [S, Y, convergence_error] = jstsp_mex('proposed_algorithm', subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type, indx_S);
  elseif nargout == 2
    [S, Y] = jstsp_mex('proposed_algorithm', subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type, indx_S);
  else
    S = jstsp_mex('proposed_algorithm', subY, Omega, A, B, Imax, tau_Y, tau_S, rho, type, indx_S);
  end
end
