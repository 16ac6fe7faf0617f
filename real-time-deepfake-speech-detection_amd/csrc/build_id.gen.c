const char afx_build_id_str[] = "5165d4fa6769";
